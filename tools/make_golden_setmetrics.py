"""Golden vectors for the set-vs-set evaluation metrics (second half): the reference's OWN numpy code
— gecco-jax/src/gecco_jax/benchmark.py:128-156, `BenchmarkCallback._assemble_dist_m`, `_one_nn_acc`, `_mmd`, `_cov` — executed from
/root/reference at run time (the methods' source is compiled as it stands; nothing is copied into the repository; the module itself
cannot be imported: jax / equinox / tensorboard are absent) on distance matrices of seeded synthetic cloud sets, and the oracle's
restatement (oracle/cpu_ref.py::set_metrics, set_pairwise_distance) checked against it.  Build container only.

  python tools/make_golden_setmetrics.py        -> tests/golden/setmetrics.npz (inputs are regenerated from seeds by the tests)
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, cpu_ref  # noqa: E402

REF = "/root/reference/gecco-jax/src/gecco_jax/benchmark.py"
METHODS = ("_assemble_dist_m", "_one_nn_acc", "_mmd", "_cov")


def reference_methods():
    """The four methods of BenchmarkCallback, compiled from the reference's file and bound to a bare object."""
    src = open(REF).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BenchmarkCallback")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS]
    assert len(fns) == len(METHODS), [f.name for f in fns]
    mod = ast.Module(body=fns, type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, REF, "exec"), ns)
    obj = types.SimpleNamespace()
    for name in METHODS:
        setattr(obj, name, types.MethodType(ns[name], obj))
    return obj


def main():
    ref = reference_methods()
    out = {}
    for name, (n, N, seed, spread) in cases.SETMETRIC_CASES.items():
        samples, data = cases.setmetric_inputs(name)
        for kind, sq in (("chamfer", False), ("chamfer_squared", True)):
            dd = cpu_ref.set_pairwise_distance(data.double(), data.double(), sq).numpy()
            ss = cpu_ref.set_pairwise_distance(samples.double(), samples.double(), sq).numpy()
            sd = cpu_ref.set_pairwise_distance(samples.double(), data.double(), sq).numpy()
            ref.dd_dist = dd
            got = {"1-nn": float(ref._one_nn_acc(ss.copy(), sd.copy())), "mmd": float(ref._mmd(sd)), "cov": float(ref._cov(sd))}
            mine = cpu_ref.set_metrics(ss, sd, dd)
            for k in got:
                assert got[k] == mine[k], (name, kind, k, got[k], mine[k])
            tag = f"{name}/{kind}"
            out[f"{tag}/ss"], out[f"{tag}/sd"], out[f"{tag}/dd"] = ss.astype(np.float32), sd.astype(np.float32), dd.astype(np.float32)
            out[f"{tag}/metrics"] = np.array([got["1-nn"], got["mmd"], got["cov"]], dtype=np.float64)
            print(f"{tag}: n {n} x N {N}: 1-NNA {got['1-nn']:.4f}  MMD {got['mmd']:.6f}  COV {got['cov']:.4f}   (oracle restatement equal)")
    path = os.path.join(ROOT, "tests", "golden", "setmetrics.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
