"""Multi-GPU execution of the denoiser: one process per GPU, batch-sharded replicas.

Every cross-element coupling of the network is *within* a sample (GroupNorm statistics per (sample, group), softmax
per (sample, head)), so evaluation, sampling and upsampling shard by batch with NO data-path collective
("replicas only").  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests) is used for rendezvous, barriers, the max-over-ranks timing and — only when the caller wants the
union on every rank — an all-gather of the finished (B, N, 3) clouds.

Sharding invariance: the HIP forward is bit-identical for a sample whatever batch it is evaluated in (no summation
order depends on B), and `sample_noise` derives each sample's noise from (seed, global sample index) alone, so a
sharded `sample_stochastic` returns exactly the clouds a single GPU would.
"""
from __future__ import annotations

import os
from typing import Callable, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def env_rank_world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str = "nccl", device: torch.device | None = None, force: bool = False) -> tuple[int, int]:
    """Initialise the default process group from the environment; no-op for world_size 1 unless `force` (a group of one
    rank: the collectives of the training step then execute on RCCL on a 1-GPU box too; needs MASTER_PORT)."""
    rank, world, _ = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` samples owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sample_noise(shape_per_sample: Sequence[int], num_draws: int, seed: int, lo: int, hi: int,
                 device: torch.device | str, dtype=torch.float32) -> Tensor:
    """(num_draws, hi - lo, *shape_per_sample) standard normal noise where sample g's draws depend only on
    (seed, g): the same bits whichever rank / shard size generates them."""
    out = torch.empty((num_draws, hi - lo, *shape_per_sample), device=device, dtype=dtype)
    for j, g in enumerate(range(lo, hi)):
        gen = torch.Generator(device=device).manual_seed((int(seed) * 1_000_003 + g) % (2 ** 63 - 1))
        out[:, j] = torch.randn((num_draws, *shape_per_sample), device=device, dtype=dtype, generator=gen)
    return out


def max_over_ranks(seconds: float, device: torch.device | str = "cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_gather_batch(local: Tensor, total: int) -> Tensor:
    """Concatenate the per-rank shards (possibly of unequal size) along dim 0 on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    m = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((m, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


class FlatGradBuffer:
    """The gradient-storage half of `gecco_amd.optim.FusedAdamEMA` on its own: one flat fp32 buffer with every
    `p.grad` a 16-byte aligned view of it, for optimizers that are not the fused one (and for the gloo tests).
    Offers what `BucketedGradAllReducer` needs: `flat_grad()`, `spans()`, `grad_scale`, `zero_grad()`, `gather_grads()`."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        off, self._spans = 0, []
        for p in self.params:
            self._spans.append((p, off, p.numel()))
            off += (p.numel() + 3) // 4 * 4
        dev = self.params[0].device if self.params else torch.device("cpu")
        self._g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0
        self._missing_ids: set[int] = set()   # as FusedAdamEMA: parameters that had no gradient when gathered (set_to_none=True)
        self._missing_grad = 0
        self.zero_grad()

    def flat_grad(self) -> Tensor:
        return self._g

    def spans(self):
        return list(self._spans)

    def zero_grad(self, set_to_none: bool = False) -> None:
        """set_to_none: p.grad = None — autograd hands its gradient tensors over instead of adding them to zeros, and
        `gather_grads` (called by the reducer per bucket) copies them into the flat buffer (FusedAdamEMA.zero_grad)."""
        if set_to_none:
            for p, _, _ in self._spans:
                p.grad = None
            return
        self._g.zero_()
        for p, o, k in self._spans:
            if p.grad is None or p.grad.data_ptr() != self._g.data_ptr() + 4 * o:
                p.grad = self._g[o:o + k].view(p.shape)

    @torch.no_grad()
    def gather_grads(self, params=None) -> None:
        span_of = {id(p): (o, k) for p, o, k in self._spans}
        src, dst = [], []
        for p in (params if params is not None else self.params):
            o, k = span_of[id(p)]
            gr = p.grad
            if gr is not None and gr.data_ptr() == self._g.data_ptr() + 4 * o:
                continue
            v = self._g[o:o + k].view(p.shape)
            if gr is None:
                v.zero_()
                if id(p) not in self._missing_ids:
                    self._missing_ids.add(id(p))
                    self._missing_grad += 1
            else:
                src.append(gr)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def take_missing(self) -> int:
        """Parameters without a gradient on ANY rank this step (after the reducer's finish()), and reset."""
        n, self._missing_grad = self._missing_grad, 0
        self._missing_ids.clear()
        return n


class BucketedGradAllReducer:
    """Gradient all-reduce of the data-parallel training step, overlapped with the backward pass.

    The gradients live in ONE flat fp32 buffer (`FusedAdamEMA.flat_grad()`: every `p.grad` is a view of it, autograd
    accumulates in place).  The buffer is cut into buckets of whole parameters (~8 MB each, 7 for the shipped model)
    and a post-accumulate hook counts a bucket's parameters as their gradients land.  Collectives are issued in ONE
    FIXED ORDER on every rank — last bucket first, the order in which the backward completes them (layers are
    differentiated last to first): a complete bucket is all-reduced (SUM, asynchronously) as soon as every bucket
    after it has been, so ranks whose hooks fire in different orders (or on which a parameter receives no gradient)
    still issue the same sequence of same-sized messages, as DDP guarantees with its bucket order.  On RCCL the
    collective runs on the communicator's own stream behind the compute stream's work so far and overlaps the rest of
    the backward.  `finish()` reduces what never completed (parameters without a gradient this step), waits for every
    handle and sets `optimizer.grad_scale = 1 / world`, which the fused Adam kernel applies while reading g: no
    averaging pass, no copy in, no copy out.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce
    of 54 MB is per-link bound at ~0.6 ms, so a handful of ~8 MB messages is the right granularity here (DDP's 25 MB
    buckets would leave the last bucket exposed; per-tensor messages would be latency-bound).

    ONE backward per `finish()`: a second `backward()` before `finish()` (gradient accumulation) would add local
    gradients onto slices that are already summed over ranks — the hook raises when that happens.

    `enabled = False`: no collective is issued anywhere (hooks, `finish()`); gradients are still gathered into the flat
    buffer and `grad_scale` is 1 — a step "without the all-reduce" for measuring what the overlap leaves exposed.
    `force_collective = True`: the collectives are issued even in a group of one rank (the RCCL path executes on a
    1-GPU box: tests/test_hip_nccl.py, `bench.py --train --force-collective`).

    Reference: Lightning's implicit DDP (example_configs/shapenet_airplane_unconditional.py:59-77), JAX `lax.pmean`
    (gecco-jax models/diffusion.py:571-573).  Works on any backend (gloo in the CPU tests).
    """

    def __init__(self, optimizer, bucket_bytes: int = 8 << 20, group=None, force_collective: bool = False):
        self.opt = optimizer
        self.group = group
        self.force_collective = force_collective
        self.flat = optimizer.flat_grad()
        spans = [(p, off, n) for p, off, n in optimizer.spans()]   # (param, offset, numel) in parameter order
        # buckets are contiguous slices of the flat buffer, closed whenever they reach bucket_bytes
        self.buckets: list[dict] = []
        cur = None
        for p, off, n in spans:
            if cur is None:
                cur = {"lo": off, "hi": off, "params": []}
            cur["params"].append(p)
            cur["hi"] = off + (n + 3) // 4 * 4
            if (cur["hi"] - cur["lo"]) * 4 >= bucket_bytes:
                self.buckets.append(cur)
                cur = None
        if cur is not None:
            self.buckets.append(cur)
        self._bucket_of = {}
        for b in self.buckets:
            b["hooked"] = [p for p in b["params"] if p.requires_grad]   # only these can ever report
            b["pending"] = len(b["hooked"])
            b["launched"] = False
            b["seen"] = set()
            for p in b["params"]:
                self._bucket_of[id(p)] = b
        self._next = len(self.buckets) - 1            # the next bucket to all-reduce (fixed order: last to first)
        self._handles: list = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready) for p, _, _ in spans if p.requires_grad]
        self.enabled = True
        self.collectives_issued = 0                   # lifetime count (tests; the bench's report)

    def world(self) -> int:
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _collective(self) -> bool:
        if not self.enabled:
            return False
        if self.force_collective:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("force_collective needs an initialised process group (distributed.init(..., force=True))")
            return True
        return self.world() > 1

    def _launch(self, b: dict) -> None:
        b["launched"] = True
        if not self._collective():
            return                                 # finish() / optimizer.step() gather whatever is not in the flat buffer yet
        from .autograd import sync_side_stream
        sync_side_stream()                         # weight gradients issued on the side stream (autograd._linear_dw)
        self.opt.gather_grads(b["params"])        # zero_grad(set_to_none=True): the bucket's gradients into their slice
        view = self.flat[b["lo"]:b["hi"]]
        self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.collectives_issued += 1

    def _flush(self, everything: bool = False) -> None:
        """Issue, in the fixed order, every bucket that is complete (or, from finish(), every bucket left)."""
        while self._next >= 0:
            b = self.buckets[self._next]
            if not (everything or b["pending"] == 0):
                return
            self._launch(b)
            self._next -= 1

    def _ready(self, p) -> None:
        if not self.enabled:
            return
        b = self._bucket_of[id(p)]
        if b["launched"] and self._collective():
            raise RuntimeError("a gradient arrived for a bucket that was already all-reduced: call finish() after every "
                               "backward() (gradient accumulation over several backward passes is not supported)")
        if id(p) in b["seen"]:
            return                                 # a second contribution to the same parameter (it is used twice)
        b["seen"].add(id(p))
        b["pending"] -= 1
        if b["pending"] == 0:
            if self.flat.data_ptr() != self.opt.flat_grad().data_ptr():
                raise RuntimeError("gradient storage moved: rebuild the BucketedGradAllReducer after the optimizer")
            self._flush()

    def finish(self) -> None:
        """Call after `loss.backward()`, before `optimizer.step()`."""
        self._flush(everything=True)
        for h in self._handles:
            h.wait()
        self._handles.clear()
        self.opt.gather_grads()                    # single process / disabled: nothing was gathered bucket by bucket
        for b in self.buckets:
            b["pending"], b["launched"] = len(b["hooked"]), False
            b["seen"].clear()
        self._next = len(self.buckets) - 1
        if self._collective() and hasattr(self.opt, "_missing_ids"):
            # A gradient missing on THIS rank arrived with the sum over the ranks — if some rank produced it.  One tiny
            # all-reduce of a per-parameter "had a gradient" mask decides it parameter by parameter: what no rank supplied stays
            # missing and the optimizer's policy applies (reference DDP with find_unused_parameters=False errors there too);
            # in a group of one rank (force_collective) nothing is forgiven.  Every rank issues it, every step: same sequence.
            spans = self.opt.spans()
            if self.opt._missing_ids:
                had = self.flat.new_tensor([0.0 if id(p) in self.opt._missing_ids else 1.0 for p, _, _ in spans])
            else:   # the common case: filled on the device — a host list would reach it through a pageable copy that is ordered behind
                # the whole backward pass on this stream and holds the host until then (2.2 ms per step measured)
                had = torch.ones(len(spans), dtype=self.flat.dtype, device=self.flat.device)
            dist.all_reduce(had, op=dist.ReduceOp.SUM, group=self.group)   # not counted in collectives_issued (the bucket count)
            # The result is READ only by a rank that misses something itself (nothing to forgive otherwise): reading it is a
            # device -> host synchronisation, and one per step keeps the host from running ahead of the device (+2.2 ms per step
            # measured on the 16-mixed step, whose launches the host otherwise issues a step ahead)
            if self.opt._missing_ids:
                still = {id(p) for (p, _, _), h in zip(spans, had.tolist()) if h == 0.0 and p.requires_grad}
                self.opt._missing_ids = still
                self.opt._missing_grad = len(still)
        self.opt.grad_scale = 1.0 / self.world() if self.enabled else 1.0

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s weights (what DDP does at construction)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def sample_stochastic_sharded(sample_fn: Callable[..., Tensor], shape: Sequence[int], num_steps: int, seed: int = 42,
                              device: torch.device | str = "cuda", gather: bool = True, **kwargs) -> Tensor:
    """Batch-sharded `Diffusion.sample_stochastic`: rank r draws the clouds [lo, hi) of the global batch `shape[0]`.
    `sample_fn(shape, noise=..., num_steps=...)` is `functools.partial(model.sample_stochastic, context=ctx_shard)`.
    Returns the local shard, or the whole batch on every rank when `gather`."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    lo, hi = shard_range(shape[0], rank, world)
    if hi == lo:   # global batch < world size: this rank owns no cloud — nothing to launch, an empty shard to gather
        local = torch.empty((0, *shape[1:]), device=device, dtype=torch.float64)   # the sampler state is fp64
    else:
        noise = sample_noise(tuple(shape[1:]), num_steps + 1, seed, lo, hi, device)
        local = sample_fn((hi - lo, *shape[1:]), noise=noise, num_steps=num_steps, **kwargs)
    return all_gather_batch(local, shape[0]) if gather else local
