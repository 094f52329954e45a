"""The collective of the data-parallel training step EXECUTED on RCCL (reference: Lightning's implicit
DDP / NCCL, example_configs/shapenet_airplane_unconditional.py:59-77; gecco-jax models/diffusion.py:571-573 `pmean`).

A 1-GPU box cannot run two RCCL ranks, but a group of ONE rank still goes through everything the 8-GPU run does on this
side of the wire: `init_process_group("nccl")`, the communicator and its stream, `dist.all_reduce(async_op=True)` on a
slice of the flat gradient buffer per bucket issued from the post-accumulate hooks during a HIP backward (with the
side-stream weight-gradient kernels synchronised first), the handles' waits, `grad_scale` into the fused Adam + EMA
kernel.  The all-reduce of one rank is the identity, so the step must equal — to the bit — the step with the reducer
disabled."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(tmp_path, force, amp=False):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / f"nccl_{int(force)}_{int(amp)}.npz")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               NCCL_DEBUG="VERSION", GECCO_PRECISION="bf16x3")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_child.py"), out, "1" if force else "0"] + (["amp"] if amp else []),
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return np.load(out), r.stdout + r.stderr


def test_one_rank_rccl_group_runs_the_bucketed_all_reduce(tmp_path):
    forced, log = _child(tmp_path, True)
    plain, _ = _child(tmp_path, False)
    print(log[-1500:])
    # RCCL's library really was in the process: NCCL_DEBUG=VERSION makes it print its version line at communicator creation
    assert any(("NCCL version" in ln) or ("RCCL version" in ln) for ln in log.splitlines()), log[-3000:]
    assert int(forced["buckets"]) >= 3
    assert int(forced["issued"]) == 2 * int(forced["buckets"])      # every bucket, both steps
    assert int(plain["issued"]) == 0
    for k in ("g0", "g1", "params", "ema"):
        assert np.isfinite(forced[k]).all()
        np.testing.assert_array_equal(forced[k], plain[k])          # sum over one rank = identity, to the bit


def test_one_rank_rccl_group_in_the_16_mixed_setting(tmp_path):
    """The same with the step under torch.autocast(float16) and a GradScaler around FusedAdamEMA (the reference's trainer setting on
    top of its DDP): scaled gradients go through the bucketed all-reduces, the optimizer unscales on the device — identical to the
    step without the collectives."""
    forced, _ = _child(tmp_path, True, amp=True)
    plain, _ = _child(tmp_path, False, amp=True)
    assert int(forced["issued"]) == 2 * int(forced["buckets"]) and int(plain["issued"]) == 0
    for k in ("g0", "g1", "params", "ema"):
        assert np.isfinite(forced[k]).all()
        np.testing.assert_array_equal(forced[k], plain[k])
